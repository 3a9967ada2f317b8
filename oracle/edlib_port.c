/*
 * oracle/edlib_port.c — TEST INFRASTRUCTURE ONLY (CPU oracle), never linked or imported by the
 * product path (ultra_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it.
 *
 * CPU restatement of what the reference obtains from the third-party call
 *     edlib.align(query, target, task="path", mode="HW", k)
 * at /root/reference/modules/classify_read_with_mams.py:99 (reached from :312, :353, :377).
 *
 * edlib itself (PyPI package "edlib", C++ core Martinsos/edlib, version UNPINNED by the reference:
 * setup.py:79-84) is absent from /root/reference and from this image, so this file restates the
 * published algorithm (Myers 1999 bit-vector / Hyyro 2003 block formulation, semi-global "HW"
 * search, reverse "SHW" pass for start locations, NW traceback for the first location).
 *
 * PARITY STATUS: distance / end locations / start locations are mathematically defined and are
 * pinned in tests against the O(mn) DP in this same file (oracle_hw_bruteforce).  The traceback
 * tie rule (prefer "up"=I, then "left"=D, then diagonal; SURVEY.md section 8c, assumption alpha) and
 * edlib's Hirschberg switch above 1 MiB of traceback state (assumption beta, NOT implemented:
 * rule alpha is applied at every size) are recalled behaviour of edlib 1.x, not checkable here:
 * the `accuracy`-dependent part of parity is "parity unpinned" against real edlib.
 *
 * Two independent implementations live here on purpose:
 *   oracle_hw_bruteforce : plain integer DP, ground truth for small inputs
 *   oracle_hw_bitvector  : 64-row block bit-vector port (bottom-padded last block, the way
 *                          edlib lays it out) - fast enough to serve as the CPU baseline.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OP_MATCH 0
#define OP_INSERT 1 /* consumes a query base only   (cigar 'I') */
#define OP_DELETE 2 /* consumes a target base only  (cigar 'D') */
#define OP_MISMATCH 3

typedef uint64_t word_t;

/* ------------------------------------------------------------------------------------------- */
/* Ground truth: plain DP                                                                      */
/* ------------------------------------------------------------------------------------------- */

/* Global (NW) traceback of q[0..m) vs t[0..L) on a full integer matrix, tie rule alpha.
 * Writes ops in forward order; returns number of ops, *n_ins = number of OP_INSERT. */
static int nw_traceback_full(const uint8_t *q, int m, const uint8_t *t, int L, int expect,
                             uint8_t *ops, int ops_cap, int *n_ins)
{
    size_t W = (size_t)L + 1;
    int *D = (int *)malloc(sizeof(int) * (size_t)(m + 1) * W);
    if (!D) return -1;
    for (int j = 0; j <= L; j++) D[j] = j;
    for (int i = 1; i <= m; i++) {
        D[(size_t)i * W] = i;
        for (int j = 1; j <= L; j++) {
            int diag = D[(size_t)(i - 1) * W + j - 1] + (q[i - 1] != t[j - 1]);
            int up = D[(size_t)(i - 1) * W + j] + 1;
            int left = D[(size_t)i * W + j - 1] + 1;
            int v = diag < up ? diag : up;
            D[(size_t)i * W + j] = v < left ? v : left;
        }
    }
    int rc = 0;
    if (D[(size_t)m * W + L] != expect) rc = -2;
    int i = m, j = L, n = 0, ins = 0;
    uint8_t *rev = (uint8_t *)malloc((size_t)m + L + 1);
    while (rc == 0 && (i > 0 || j > 0)) {
        int cur = D[(size_t)i * W + j];
        if (i > 0 && D[(size_t)(i - 1) * W + j] + 1 == cur) {
            rev[n++] = OP_INSERT; ins++; i--;
        } else if (j > 0 && D[(size_t)i * W + j - 1] + 1 == cur) {
            rev[n++] = OP_DELETE; j--;
        } else {
            rev[n++] = (D[(size_t)(i - 1) * W + j - 1] == cur) ? OP_MATCH : OP_MISMATCH;
            i--; j--;
        }
    }
    if (rc == 0) {
        if (ops) {
            if (n > ops_cap) rc = -3;
            else for (int a = 0; a < n; a++) ops[a] = rev[n - 1 - a];
        }
        *n_ins = ins;
    }
    free(rev);
    free(D);
    return rc == 0 ? n : rc;
}

/* Semi-global search of q in t by plain DP.
 * ed = min_j D[m][j] (j = 1..n), -1 when > k.  Locations: every arg-min column as inclusive
 * 0-based (start, end); start = smallest s with NW(q, t[s..end]) == ed.
 * Returns 0, or <0 on error (-3: location/ops capacity too small, n_loc then holds the need). */
int oracle_hw_bruteforce(const uint8_t *q, int m, const uint8_t *t, int n, int k,
                         int *ed, int *n_loc, int *starts, int *ends, int cap,
                         int *first_ins, uint8_t *ops, int ops_cap, int *ops_len)
{
    *ed = -1; *n_loc = 0; *first_ins = 0; if (ops_len) *ops_len = 0;
    if (m <= 0 || n <= 0) return 0; /* assumption delta: unreachable from the reference's call sites */
    int *prev = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int *cur = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int j = 0; j <= n; j++) prev[j] = 0;
    for (int i = 1; i <= m; i++) {
        cur[0] = i;
        for (int j = 1; j <= n; j++) {
            int v = prev[j - 1] + (q[i - 1] != t[j - 1]);
            if (prev[j] + 1 < v) v = prev[j] + 1;
            if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1;
            cur[j] = v;
        }
        int *tmp = prev; prev = cur; cur = tmp;
    }
    int best = prev[1];
    for (int j = 2; j <= n; j++) if (prev[j] < best) best = prev[j];
    if (k >= 0 && best > k) { free(prev); free(cur); return 0; }
    int cnt = 0;
    for (int j = 1; j <= n; j++) if (prev[j] == best) { if (cnt < cap) ends[cnt] = j - 1; cnt++; }
    free(prev); free(cur);
    *ed = best; *n_loc = cnt;
    if (cnt > cap) return -3;
    /* starts: anchored-start DP of reversed query against the reversed target prefix */
    for (int a = 0; a < cnt; a++) {
        int e = ends[a];
        int lim = m + best; if (lim > e + 1) lim = e + 1;
        int *p = (int *)malloc(sizeof(int) * (size_t)(lim + 1));
        int *c = (int *)malloc(sizeof(int) * (size_t)(lim + 1));
        for (int j = 0; j <= lim; j++) p[j] = j;
        for (int i = 1; i <= m; i++) {
            c[0] = i;
            uint8_t qc = q[m - i];
            for (int j = 1; j <= lim; j++) {
                int v = p[j - 1] + (qc != t[e - (j - 1)]);
                if (p[j] + 1 < v) v = p[j] + 1;
                if (c[j - 1] + 1 < v) v = c[j - 1] + 1;
                c[j] = v;
            }
            int *tmp = p; p = c; c = tmp;
        }
        int col = -1;
        for (int j = 1; j <= lim; j++) if (p[j] == best) col = j;
        free(p); free(c);
        if (col < 0) return -4;
        starts[a] = e - col + 1;
    }
    int L = ends[0] - starts[0] + 1;
    int r = nw_traceback_full(q, m, t + starts[0], L, best, ops, ops_cap, first_ins);
    if (r < 0) return r;
    if (ops_len) *ops_len = r;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Bit-vector port (block formulation, bottom-padded last block)                               */
/* ------------------------------------------------------------------------------------------- */

/* One 64-row block, one column.  hin/hout in {-1,0,+1}. */
static inline int block_step(word_t *Pv, word_t *Mv, word_t Eq, int hin)
{
    word_t pv = *Pv, mv = *Mv;
    word_t hneg = hin < 0 ? 1u : 0u, hpos = hin > 0 ? 1u : 0u;
    word_t Xv = Eq | mv;
    Eq |= hneg;
    word_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
    word_t Ph = mv | ~(Xh | pv);
    word_t Mh = pv & Xh;
    int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
    Ph = (Ph << 1) | hpos;
    Mh = (Mh << 1) | hneg;
    *Pv = Mh | ~(Xv | Ph);
    *Mv = Ph & Xv;
    return hout;
}

/* Peq[sym*B + b]: bit r set iff q[64b + r] == sym; padding rows of the last block match all. */
static word_t *build_peq(const uint8_t *q, int m, int B, int reversed)
{
    word_t *peq = (word_t *)calloc((size_t)256 * B, sizeof(word_t));
    if (!peq) return NULL;
    for (int i = 0; i < m; i++) {
        uint8_t ch = reversed ? q[m - 1 - i] : q[i];
        peq[(size_t)ch * B + (i >> 6)] |= (word_t)1 << (i & 63);
    }
    int W = B * 64 - m;
    if (W > 0) {
        word_t pad = ~(word_t)0 << (64 - W);
        for (int s = 0; s < 256; s++) peq[(size_t)s * B + (B - 1)] |= pad;
    }
    return peq;
}

typedef struct { int best; int cnt; int last_pos; } search_res;

/* Column sweep.  top_hin = 0: free start in the target (HW); +1: anchored start (SHW).
 * tstep = +1 / -1 walks the target forwards / backwards from t0.
 * Collects every column whose row-m score equals the running best (<= k); positions are real
 * 0-based column indices (the padded bottom row lags W columns behind). */
static search_res sweep(const word_t *peq, int B, int m, const uint8_t *t0, int tstep, int n, int k,
                        int top_hin, int *pos, int cap)
{
    search_res R; R.best = -1; R.cnt = 0; R.last_pos = -1;
    int W = B * 64 - m;
    word_t *Pv = (word_t *)malloc(sizeof(word_t) * B), *Mv = (word_t *)malloc(sizeof(word_t) * B);
    int *score = (int *)malloc(sizeof(int) * B);
    for (int b = 0; b < B; b++) { Pv[b] = ~(word_t)0; Mv[b] = 0; score[b] = 64 * (b + 1); }
    int lim = k; /* only scores <= lim are collected; lim tightens to the best seen */
    for (int c = 0; c < n; c++) {
        uint8_t ch = t0[(ptrdiff_t)c * tstep];
        int h = top_hin;
        for (int b = 0; b < B; b++) {
            h = block_step(&Pv[b], &Mv[b], peq[(size_t)ch * B + b], h);
            score[b] += h;
        }
        int s = score[B - 1], p = c - W;
        if (p >= 0 && s <= lim) {
            if (s != R.best) { R.cnt = 0; R.best = s; lim = s; }
            if (pos && R.cnt < cap) pos[R.cnt] = p;
            R.cnt++; R.last_pos = p;
        }
    }
    /* the last W real columns sit above the bottom row of the last block in the final column */
    int s = score[B - 1];
    for (int r = 63, x = W; x > 0; r--) {
        /* moving one row up from the cell with score s */
        s -= (int)((Pv[B - 1] >> r) & 1) - (int)((Mv[B - 1] >> r) & 1);
        x--;
        int p = n - 1 - x; /* row m + x holds real column n - x */
        if (p >= 0 && s <= lim) {
            if (s != R.best) { R.cnt = 0; R.best = s; lim = s; }
            if (pos && R.cnt < cap) pos[R.cnt] = p;
            R.cnt++; R.last_pos = p;
        }
    }
    free(Pv); free(Mv); free(score);
    return R;
}

/* NW of q vs t[0..L) storing every column, then traceback with tie rule alpha. */
static int nw_traceback_bits(const uint8_t *q, int m, const uint8_t *t, int L, int expect,
                             uint8_t *ops, int ops_cap, int *n_ins)
{
    int B = (m + 63) / 64;
    word_t *peq = build_peq(q, m, B, 0);
    size_t cells = (size_t)B * L;
    word_t *P = (word_t *)malloc(sizeof(word_t) * cells), *M = (word_t *)malloc(sizeof(word_t) * cells);
    int *S = (int *)malloc(sizeof(int) * cells);
    word_t *Pv = (word_t *)malloc(sizeof(word_t) * B), *Mv = (word_t *)malloc(sizeof(word_t) * B);
    int *score = (int *)malloc(sizeof(int) * B);
    for (int b = 0; b < B; b++) { Pv[b] = ~(word_t)0; Mv[b] = 0; score[b] = 64 * (b + 1); }
    for (int c = 0; c < L; c++) {
        int h = 1;
        for (int b = 0; b < B; b++) {
            h = block_step(&Pv[b], &Mv[b], peq[(size_t)t[c] * B + b], h);
            score[b] += h;
            P[(size_t)c * B + b] = Pv[b]; M[(size_t)c * B + b] = Mv[b]; S[(size_t)c * B + b] = score[b];
        }
    }
    free(Pv); free(Mv); free(score); free(peq);
#define CELL(i, c) /* D[i+1][c+1] for 0-based row i, column c; boundaries handled by caller */ \
    (S[(size_t)(c) * B + ((i) >> 6)] \
     - (((i) & 63) == 63 ? 0 : (__builtin_popcountll(P[(size_t)(c) * B + ((i) >> 6)] >> (((i) & 63) + 1)) \
                               - __builtin_popcountll(M[(size_t)(c) * B + ((i) >> 6)] >> (((i) & 63) + 1)))))
    int rc = 0;
    int i = m - 1, c = L - 1, n = 0, ins = 0;
    if (CELL(i, c) != expect) rc = -2;
    uint8_t *rev = (uint8_t *)malloc((size_t)m + L + 1);
    while (rc == 0 && (i >= 0 || c >= 0)) {
        int cur = (i < 0) ? c + 1 : (c < 0) ? i + 1 : CELL(i, c);
        int up = (i < 0) ? -1 : (i == 0) ? c + 1 : (c < 0) ? i : CELL(i - 1, c);
        int left = (c < 0) ? -1 : (c == 0) ? i + 1 : (i < 0) ? c : CELL(i, c - 1);
        if (i >= 0 && up + 1 == cur) { rev[n++] = OP_INSERT; ins++; i--; }
        else if (c >= 0 && left + 1 == cur) { rev[n++] = OP_DELETE; c--; }
        else {
            int ul = (i == 0) ? c : (c == 0) ? i : CELL(i - 1, c - 1);
            rev[n++] = (ul == cur) ? OP_MATCH : OP_MISMATCH; i--; c--;
        }
    }
#undef CELL
    if (rc == 0) {
        if (ops) {
            if (n > ops_cap) rc = -3;
            else for (int a = 0; a < n; a++) ops[a] = rev[n - 1 - a];
        }
        *n_ins = ins;
    }
    free(rev); free(P); free(M); free(S);
    return rc == 0 ? n : rc;
}

/* ------------------------------------------------------------------------------------------- */
/* Exposure of the unpinned tie rule: insertion counts over ALL optimal paths                   */
/* ------------------------------------------------------------------------------------------- */

/* Global alignments of q[0..m) with t[0..L) of cost `expect` (the first location of a hit): the smallest and the
 * largest number of inserted query bases ('I' columns, "up" moves) over every optimal path - whatever tie rule, or
 * Hirschberg split, a real edlib applies, its path's count lies in [*imin, *imax], and accuracy = (L - ED + I) / (L + I)
 * grows with I.  Plain DP over the DAG of tight predecessors, restricted to |row - column| <= expect (no optimal
 * path leaves that band).  Returns 0, -2 when the distance is not `expect`. */
int oracle_ins_range(const uint8_t *q, int m, const uint8_t *t, int L, int expect, int *imin, int *imax)
{
    const int BIG = 1 << 29;
    const int W = 2 * expect + 3;              /* band slots: column j of row i at slot j - i + expect + 1 */
    int *D0 = (int *)malloc(sizeof(int) * W * 6);
    if (!D0) return -1;
    int *D1 = D0 + W, *lo0 = D1 + W, *lo1 = lo0 + W, *hi0 = lo1 + W, *hi1 = hi0 + W;
    for (int s = 0; s < W; s++) { D0[s] = BIG; lo0[s] = 0; hi0[s] = 0; }
    for (int j = 0; j <= L && j <= expect; j++) { D0[j + expect + 1] = j; lo0[j + expect + 1] = 0; hi0[j + expect + 1] = 0; }
    for (int i = 1; i <= m; i++) {
        for (int s = 0; s < W; s++) { D1[s] = BIG; lo1[s] = 0; hi1[s] = 0; }
        int jlo = i - expect < 0 ? 0 : i - expect, jhi = i + expect > L ? L : i + expect;
        for (int j = jlo; j <= jhi; j++) {
            const int s = j - i + expect + 1;   /* this cell; (i-1, j) sits at s + 1 of the previous row, (i-1, j-1) at s */
            const int up = D0[s + 1] < BIG ? D0[s + 1] + 1 : BIG;
            const int diag = (j > 0 && D0[s] < BIG) ? D0[s] + (q[i - 1] != t[j - 1]) : BIG;
            const int left = (j > jlo && D1[s - 1] < BIG) ? D1[s - 1] + 1 : BIG;
            int v = up < diag ? up : diag;
            if (left < v) v = left;
            if (v >= BIG) continue;
            int l = BIG, h = -1;
            if (up == v) { if (lo0[s + 1] + 1 < l) l = lo0[s + 1] + 1; if (hi0[s + 1] + 1 > h) h = hi0[s + 1] + 1; }
            if (diag == v) { if (lo0[s] < l) l = lo0[s]; if (hi0[s] > h) h = hi0[s]; }
            if (left == v) { if (lo1[s - 1] < l) l = lo1[s - 1]; if (hi1[s - 1] > h) h = hi1[s - 1]; }
            D1[s] = v; lo1[s] = l; hi1[s] = h;
        }
        int *x;
        x = D0; D0 = D1; D1 = x; x = lo0; lo0 = lo1; lo1 = x; x = hi0; hi0 = hi1; hi1 = x;
    }
    int rc = 0;
    const int s = L - m + expect + 1;
    if (s < 0 || s >= W || D0[s] != expect) rc = -2;
    else { *imin = lo0[s]; *imax = hi0[s]; }
    {   /* D0 .. hi1 were swapped around: free the block's base address */
        int *base = D0;
        if (D1 < base) base = D1;
        if (lo0 < base) base = lo0;
        if (lo1 < base) base = lo1;
        if (hi0 < base) base = hi0;
        if (hi1 < base) base = hi1;
        free(base);
    }
    return rc;
}

/* CPU-BASELINE switch only (bench.py's cpu_baseline leg sets it; every test leaves it 0, so that the checker computes
 * every cell): 1 = answer -1 at once when the query is more than k longer than the target, which is what edlib's band
 * amounts to for those pairs and what libultra_hip does at stage time - a baseline without it would be charged for
 * cells neither edlib nor the GPU path computes. */
int oracle_length_prefilter = 0;
void oracle_set_length_prefilter(int on) { oracle_length_prefilter = on; }

/* Same contract as oracle_hw_bruteforce. want_path = 0 skips the traceback (first_ins = 0). */
int oracle_hw_bitvector(const uint8_t *q, int m, const uint8_t *t, int n, int k,
                        int *ed, int *n_loc, int *starts, int *ends, int cap,
                        int *first_ins, uint8_t *ops, int ops_cap, int *ops_len, int want_path)
{
    *ed = -1; *n_loc = 0; *first_ins = 0; if (ops_len) *ops_len = 0;
    if (m <= 0 || n <= 0) return 0;
    int B = (m + 63) / 64;
    if (k < 0 || k > m) k = m; /* a semi-global hit never costs more than the whole query */
    if (oracle_length_prefilter && m - n > k) return 0;
    word_t *peq = build_peq(q, m, B, 0);
    search_res F = sweep(peq, B, m, t, +1, n, k, 0, ends, cap);
    free(peq);
    if (F.best < 0) return 0;
    *ed = F.best; *n_loc = F.cnt;
    if (F.cnt > cap) return -3;
    word_t *rpeq = build_peq(q, m, B, 1);
    for (int a = 0; a < F.cnt; a++) {
        int e = ends[a];
        int lim = m + F.best; if (lim > e + 1) lim = e + 1;
        search_res Rv = sweep(rpeq, B, m, t + e, -1, lim, F.best, 1, NULL, 0);
        if (Rv.best != F.best) { free(rpeq); return -4; }
        starts[a] = e - Rv.last_pos;
    }
    free(rpeq);
    if (want_path) {
        int L = ends[0] - starts[0] + 1;
        int r = nw_traceback_bits(q, m, t + starts[0], L, F.best, ops, ops_cap, first_ins);
        if (r < 0) return r;
        if (ops_len) *ops_len = r;
    }
    return 0;
}

/* Forward pass only (distance + number of end columns): the unit the GCUPS figure counts. */
int oracle_hw_forward_only(const uint8_t *q, int m, const uint8_t *t, int n, int k, int *ed, int *n_loc)
{
    *ed = -1; *n_loc = 0;
    if (m <= 0 || n <= 0) return 0;
    int B = (m + 63) / 64;
    if (k < 0 || k > m) k = m;
    word_t *peq = build_peq(q, m, B, 0);
    search_res F = sweep(peq, B, m, t, +1, n, k, 0, NULL, 0);
    free(peq);
    if (F.best >= 0) { *ed = F.best; *n_loc = F.cnt; }
    return 0;
}

/* Batched driver used by tests and by bench.py's cpu_baseline leg (single thread).
 * arena: all sequences; per pair q_off/q_len/t_off/t_len/k.  Locations go to a CSR whose
 * per-pair capacity is loc_off[p+1]-loc_off[p]. Returns number of pairs that overflowed. */
int oracle_hw_batch(const uint8_t *arena, int n_pairs,
                    const int64_t *q_off, const int32_t *q_len, const int64_t *t_off, const int32_t *t_len,
                    const int32_t *k, const int64_t *loc_off,
                    int32_t *ed, int32_t *n_loc, int32_t *starts, int32_t *ends, int32_t *first_ins,
                    int want_path)
{
    int overflow = 0;
    for (int p = 0; p < n_pairs; p++) {
        int cap = (int)(loc_off[p + 1] - loc_off[p]);
        int e, nl, fi;
        int rc = oracle_hw_bitvector(arena + q_off[p], q_len[p], arena + t_off[p], t_len[p], k[p],
                                     &e, &nl, starts + loc_off[p], ends + loc_off[p], cap, &fi, NULL, 0, NULL,
                                     want_path);
        ed[p] = e; n_loc[p] = nl; first_ins[p] = fi;
        if (rc == -3) overflow++;
        else if (rc < 0) return rc;
    }
    return overflow;
}
